// Exercises the C++ shims the way the reference's callers use the original classes
// (CirclesEventFrame.cpp:66-72 for DBSCAN; eventCameraCalib.cpp:138-163,49-56 for the stream,
// container and frames).  Built and run by tests/test_gpu_shims.py on the GPU box.
//   usage: test_shims events.bin
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../eventcalib_amd/csrc/host/multi_process.hpp"

#define CHECK(c)                                                        \
    do {                                                                \
        if (!(c)) {                                                     \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                   \
        }                                                               \
    } while (0)

int main(int argc, char **argv) {
    using namespace opengv2;
    // --- DBSCAN: the survey's 3-point demo (order dependent pruning quirk) ---
    {
        std::vector<Vector2d> v = {{{4, 5}}, {{4, 0}}, {{0, 0}}};
        DBSCAN<Vector2d, double> db;
        CHECK(db.Run(&v, 2, 4.0, 1) == 0);
        CHECK(db.Clusters.size() == 1 && db.Clusters[0].size() == 1 && db.Clusters[0][0] == 1);
        CHECK(db.Noise.size() == 2 && db.Noise[0] == 0 && db.Noise[1] == 2);
        std::vector<Vector2d> w = {{{4, 0}}, {{0, 0}}, {{4, 5}}};
        CHECK(db.Run(&w, 2, 4.0, 1) == 0);
        CHECK(db.Clusters.size() == 1 && db.Clusters[0].size() == 2 && db.Noise.size() == 1 && db.Noise[0] == 2);
        std::vector<Vector2d> empty;
        CHECK(db.Run(&empty, 2, 4.0, 1) == 1);   // FAILED, dbscan.h:121
        CHECK(db.Run(&v, 2, 4.0, 0) == 1);       // FAILED, dbscan.h:123
        CHECK(db.Run(&v, 0, 4.0, 1) == 1);       // FAILED, dbscan.h:122
    }
    if (argc < 2) {
        std::printf("shims ok (dbscan only)\n");
        return 0;
    }
    // --- stream -> container -> frames, as the reference driver does ---
    EventStream es(argv[1]);
    auto container = std::make_shared<EventContainer>();
    while (!es.isEnd()) {
        container->emplace(es.current());
        es.next();
    }
    es.close();
    CHECK(container->size() > 1000);
    bool threw = false;
    try {
        EventStream missing("/nonexistent/file.bin");
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    CHECK(threw);
    const double t0 = container->firstTime();
    auto pattern = std::make_shared<CirclePatternParameters>();
    CirclesEventFrame frame(container, {t0, t0 + 1.5e-3}, pattern);
    CHECK(frame.circleRadiusThreshold() == 15.511363636363637);
    const bool ok = frame.extractFeatures();
    const FrameDetection &d = frame.detection();
    CHECK(frame.eventsNum() == (int) (d.positive.size() + d.negative.size()));
    CHECK(d.positive.size() > 100 && d.negative.size() > 100);
    // the DBSCAN shim on the frame's pixel sets must reproduce the pipeline's labels
    DBSCAN<Vector2d, double> db;
    std::vector<Vector2d> pos = d.positive;
    CHECK(db.Run(&pos, 2, 4.0, 2) == 0);
    CHECK(db.Clusters.size() == d.nClustersPos);
    for (size_t c = 0; c < db.Clusters.size(); c++)
        for (uint pid : db.Clusters[c]) CHECK(d.labelsPos[pid] == (int32_t) c);
    for (uint pid : db.Noise) CHECK(d.labelsPos[pid] == -1);
    std::printf("shims ok: %zu events, window: %zu + / %zu - pixels, %u/%u clusters, %zu candidates, extract=%d\n",
                container->size(), d.positive.size(), d.negative.size(), d.nClustersPos, d.nClustersNeg,
                d.candidates.size(), (int) ok);
    // batch of overlapping windows through detect_windows
    std::vector<std::pair<double, double>> wins;
    for (int k = 0; k < 6; k++) wins.push_back({t0 + k * 5e-4, t0 + k * 5e-4 + 1.5e-3});
    std::vector<FrameDetection> out;
    ecal_detect_params prm;
    prm.dbscan_eps = 4;
    prm.dbscan_min_samples = 2;
    prm.cluster_min_sample = 5;
    prm.need_clusters = 36;
    prm.circle_radius_threshold = frame.circleRadiusThreshold();
    prm.fit_circle = 0;
    prm.knn_num = 3;
    prm.rows = 9;
    prm.cols = 4;
    detect_windows(*container, wins, prm, out);
    CHECK(out.size() == 6 && out[0].positive.size() == d.positive.size());
    if (ok) {  // ordered features: 36 circles, neighbours in a row are ~2 squares apart and ordered consistently
        CHECK(frame.features().size() == 36);
        CHECK(d.gridFound && d.orderIdxs.size() == 36);
    }
    // rectifyFeatures (CirclesEventFrame.cpp:417-638) with the pose of the window's mid time (argv[2]: rows of
    // t + Rcw row-major + tcw) and with a pose 5 cm off
    if (ok && argc > 2) {
        std::ifstream pf(argv[2], std::ios::binary);
        std::vector<double> row(13), best(13);
        double bd = 1e300;
        const double tm = t0 + 0.75e-3;
        while (pf.read(reinterpret_cast<char *>(row.data()), 13 * sizeof(double)))
            if (std::fabs(row[0] - tm) < bd) bd = std::fabs(row[0] - tm), best = row;
        CHECK(bd < 1e-3);
        double R[9], tc[3], tbad[3];
        for (int i = 0; i < 9; i++) R[i] = best[1 + i];
        for (int i = 0; i < 3; i++) tc[i] = best[10 + i], tbad[i] = best[10 + i] + (i == 0 ? 5.0 : 0.0);
        const CirclesEventFrame::Camera cam{359.67525, 359.67525, 172.5, 129.5, {-0.34991902, -0.014698517, 0, 0, 0.59684463}};
        CirclesEventFrame good(container, {t0, t0 + 1.5e-3}, pattern), bad(container, {t0, t0 + 1.5e-3}, pattern);
        CHECK(good.extractFeatures() && bad.extractFeatures());
        const std::vector<CirclesEventFrame::CalibCircle> before = good.features();
        CHECK(good.rectifyFeatures({}, R, tc, cam));
        CHECK(good.features().size() >= 30 && good.features().size() == good.featureLandmark().size());
        for (size_t i = 0; i < good.features().size(); i++) {   // refit centres stay near the midpoint candidates
            const auto &a = good.features()[i];
            const auto &b = before[(size_t) good.featureLandmark()[i]];
            CHECK(std::hypot(a.location[0] - b.location[0], a.location[1] - b.location[1]) < 12.0);
        }
        CHECK(!bad.rectifyFeatures({}, R, tbad, cam));
        std::printf("rectify: %zu of 36 features kept with the true pose, frame rejected with the shifted pose\n",
                    good.features().size());
    }
    // the driver's adaptive windowing + keyframe gate (eventCameraCalib.cpp:34-97, EventCalibIni.cpp:23-97)
    CirclesEventFrame::Params fp;
    const double step = 5e-4;
    std::vector<KeyFrame> kfs = detect_keyframes(*container, pattern, fp, step, 4000, 4, t0, container->lastTime());
    std::printf("adaptive windowing: %zu keyframes\n", kfs.size());
    CHECK(kfs.size() >= 2);
    for (size_t k = 0; k < kfs.size(); k++) {
        CHECK(kfs[k].features.size() == 36);
        CHECK(kfs[k].duration.second - kfs[k].duration.first >= 3 * step - 1e-12);
        CHECK(kfs[k].duration.second - kfs[k].duration.first <= 10 * step + 1e-9);
        if (k) CHECK(kfs[k].timeStamp > kfs[k - 1].timeStamp);
        std::printf("kf %.12f %.12f %d %.9f %.9f\n", kfs[k].duration.first, kfs[k].duration.second, kfs[k].eventsNum,
                    kfs[k].features[0].location[0], kfs[k].features[35].radius);
    }
    // the same search (the own-piece gate of detect_keyframes) with the policy on the device: identical keyframes
    std::vector<KeyFrame> kfd = detect_keyframes_device(*container, pattern, fp, step, 4000, 4, t0, container->lastTime(), ECAL_GATE_OWN_PIECE);
    CHECK(kfd.size() == kfs.size());
    for (size_t k = 0; k < kfd.size() && k < kfs.size(); k++) {
        CHECK(kfd[k].timeStamp == kfs[k].timeStamp && kfd[k].duration == kfs[k].duration && kfd[k].eventsNum == kfs[k].eventsNum);
        for (int c = 0; c < 36; c++)
            CHECK(kfd[k].features[c].location[0] == kfs[k].features[c].location[0] &&
                  kfd[k].features[c].location[1] == kfs[k].features[c].location[1] && kfd[k].features[c].radius == kfs[k].features[c].radius);
    }
    std::printf("device policy: %zu keyframes, identical\n", kfd.size());
    // the shim's default: the reference's single-worker gate (one keyframe map for all pieces) — only the run's very first
    // success is ungated; its first keyframe in time is the own-piece run's
    std::vector<KeyFrame> kfm = detect_keyframes_device(*container, pattern, fp, step, 4000, 4, t0, container->lastTime());
    CHECK(!kfm.empty() && kfm[0].timeStamp == kfs[0].timeStamp);
    std::printf("shared-map gate (default): %zu keyframes\n", kfm.size());
    return 0;
}
